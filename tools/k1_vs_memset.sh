# K1 (UR10, 1e6 samples) against hipMemset of the same 4.03 GB on THIS box: is the 0.71-0.84 ms box-to-box spread of K1 the
# kernel's or the box's?
python tools/hbm_write_ceiling.py 2>&1 | head -1
python bench.py --no-cpu-baseline --steps 20 --warmup 3 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('K1 %.3f ms  TSQR %.3f ms  step %.3f ms' % (k['regressor_chain']['avg_ms'], k['tsqr']['avg_ms'], d['ms_per_step']))"
rocm-smi --showclocks 2>/dev/null | grep -i "mclk\|sclk\|fclk" | head -4
