set -u
O=gpurun_out/r06_lb; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 600 python bench.py --steps 5 --warmup 2 --no-search --no-cpu-baseline --no-parity > $O/bench.json 2>$O/err.log; echo rc=$?
python -c "
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print(d['value'], json.dumps(d.get('long_passages')))
"; tail -3 $O/err.log | cut -c1-300
