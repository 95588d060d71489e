for d in 64 16 0; do echo DIRECT=$d; EFFDET_DW_WGRAD_DIRECT=$d DW_ONLY=wgrad timeout 300 python tools/dw_bench2.py 2>&1 | grep -v amdgpu | tail -4; done
