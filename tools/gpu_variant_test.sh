#!/usr/bin/env bash
# GPU box: build ONE -D variant of the library, run the parity subset against it, then bench it.  usage: tools/gpu_variant_test.sh "<-D flags>"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r2
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP -fPIC -shared $1 -o /tmp/libvar.so pymasc_amd/csrc/*.hip 2>/dev/null || { echo "build failed"; exit 1; }
PYMASC_AMD_LIB=/tmp/libvar.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
for mode in both; do
  PYMASC_AMD_LIB=/tmp/libvar.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --mode $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$mode', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
done
