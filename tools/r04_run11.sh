for spec in "2 1" "2 0" "2 1" "2 0"; do set -- $spec
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --emulate-world $1 --emulate-rank $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$1 r=$2', round(d['ms_per_step'],3), {k:(v['batches'],round(v['avg_ms'],3)) for k,v in d['scan_launches'].items()}, 'hash', round(d['rank0_ms']['hash'],3))"
done
python3 - <<'P'
import sys
sys.path.insert(0,'.')
from phylign_amd import workload as W
shapes=W.select('config3')
parts=W.assign_batches(shapes,2,capacity_bytes=int(309e9*0.85))
for r,p in enumerate(parts):
    print('rank',r,[(shapes[i].batch[:14],shapes[i].n_docs,round(shapes[i].index_bytes/1e9,1)) for i in p if shapes[i].row_bytes>256])
P
