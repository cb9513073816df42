for cfg in "16 16" "20 20" "24 20" "24 24" "32 24"; do set -- $cfg; GPU_MAX_HW_QUEUES=$1 python bench.py --streams $2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > /tmp/s.json; python - $1 $2 <<'PY'
import json,sys
d=json.load(open('/tmp/s.json')); print("hwq",sys.argv[1],"streams",sys.argv[2],round(d["ms_per_proof"],3),"ms/proof", flush=True)
PY
done
