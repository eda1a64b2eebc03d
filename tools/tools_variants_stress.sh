R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP -fPIC -shared -D$v -o /tmp/libabl.so pymasc_amd/csrc/*.hip 2>/dev/null
  for mode in both ncc; do
    PYMASC_AMD_LIB=/tmp/libabl.so python bench.py --workload stress --steps 3 --warmup 1 --no-cpu-baseline --mode $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$mode', round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
  done
done
