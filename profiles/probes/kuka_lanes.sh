for l in 64 32 16 8; do
  SRLHIP_KUKA_LANES=$l timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('lanes', $l, '%.3e' % d['value'])"
done
