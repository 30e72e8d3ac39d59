for i in 1 2; do
for v in "MVK_SMALL_BWD_UNITS=1" "MVK_SMALL_BWD_OCC=3" "MVK_LIB_PATH=$PWD/multivae_amd/libmvk_old.so"; do
env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v'[:24],d['value'],d['ms_per_step'],d['roofline_image']['image_layer_bwd']['avg_launch_us'])"
done; done
