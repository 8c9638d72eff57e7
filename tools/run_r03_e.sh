timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_render_gpu.py -m gpu -q -x 2>&1 | tail -3
for v in 1 0; do echo "== SDN_FIELD_SINGLE_KERNEL=$v"; SDN_FIELD_SINGLE_KERNEL=$v timeout 300 python tools/bench_field.py 6 2>&1 | grep pose; done
