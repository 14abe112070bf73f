cd /root/repo
timeout 200 python scripts/ab_queue.py 1024 "1,1,4" datagen,text,silesia 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['shape'], 'parse', d['parse_ms'], 'entropy', d['entropy_ms'], d['GBps'], d['sha'][:8])
    except Exception: print(l.strip()[:200])"
timeout 600 python -m pytest tests/test_gpu_compress.py tests/test_gpu_dict.py tests/test_gpu_frames.py tests/test_gpu_tables.py -m gpu -q -x 2>&1 | tail -3
python bench.py --leg records_zdict_level3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('records leg:', d['value'], d['pipeline'], d['parity']['full_size']['sha256_equals_reference_stream'])"
python bench.py --leg silesia64_level3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('silesia64 leg:', d['value'], d['pipeline'], d['parity']['full_size']['sha256_equals_reference_stream'])"
