python -m pytest tests/test_gpu_raster.py tests/test_gpu_chain.py tests/test_gpu_warp.py -m gpu -x -q --no-header 2>&1 | tail -3
python bench.py --roofline-only --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print('%-58s %7.1f %7.1f'%(k,v['ms']*1000,v['ms_cache_warm']*1000))
"
bash scripts/fwd_stage_insts.sh | head -3
python tests/fuzz_parity.py 150 2>&1 | tail -3
