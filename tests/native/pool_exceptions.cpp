// Host-side check (no GPU): an exception thrown by a parallel_for item -- on a pool thread or on the caller's --
// is handed to the caller after every item has run, the pool keeps working, and PM_GUARD_END's on_exception()
// turns it into an error code + message.  Linked against the library's object files (parallel_for is not exported).
#include "../../phylign_amd/csrc/pm_host.h"

static int guarded(int what) try {
    std::vector<int> seen(256, 0);
    parallel_for(256, [&](size_t i) {
        seen[i] = 1;
        if (what == 1 && i % 7 == 3) throw std::bad_alloc();
        if (what == 2 && i == 200) throw std::runtime_error("boom");
    });
    return PM_OK;
} PM_GUARD_END

int main() {
    for (int round = 0; round < 20; ++round) {
        std::atomic<int> ran{0};
        bool caught = false;
        try {
            parallel_for(64, [&](size_t i) { ran++; if (i == 13 || i == 40) throw std::bad_alloc(); });
        } catch (const std::bad_alloc&) { caught = true; }
        if (!caught) return 1;
        if (ran.load() != 64) return 2;                      // the other items still ran: nobody is left waiting
        std::atomic<long> sum{0};
        parallel_for(1000, [&](size_t i) { sum += (long)i; });   // the pool is intact
        if (sum.load() != 999L * 1000 / 2) return 3;
    }
    if (guarded(0) != PM_OK) return 4;
    if (guarded(1) != PM_ENOMEM || !strstr(pm_last_error(), "out of host memory")) return 5;
    if (guarded(2) != PM_EINVAL || !strstr(pm_last_error(), "boom")) return 6;
    printf("pool exceptions ok\n");
    return 0;
}
