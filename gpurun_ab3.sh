cd /root/repo
for W in "silesia --copies 8" "datagen --mib 1024" "text --mib 1024"; do for P in 0 1 2; do
  echo -n "L3 $W ZHIP_DF_PERSIST=$P: "
  ZHIP_DF_PERSIST=$P timeout 200 python bench.py --workload $W --level 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-pipelined-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['pipeline']['parse_ms'], d['pipeline']['entropy_ms'], d['parity'].get('full_size',{}).get('sha256_equals_reference_stream'), d['parity']['bytes_identical_to_oracle_first_64_units'])"
done; done
