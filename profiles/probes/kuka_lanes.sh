# Kuka rollout: envs per wavefront x workgroups per CU (run on the GPU box from the repo root)
for cfg in "64 0" "64 100" "32 100" "16 100" "32 0"; do
  set -- $cfg
  SRLHIP_KUKA_LANES=$1 SRLHIP_KUKA_LDS_KB=$2 timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('lanes', $1, 'lds_kb', $2, '%.3e' % d['value'])"
done
