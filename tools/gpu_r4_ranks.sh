#!/usr/bin/env bash
# GPU box: rehearsal of the N-rank bench on ONE GPU (gloo backend, ranks share the device): checks the code path and the
# per-rank keys of the line, not performance.   usage: tools/gpu_r4_ranks.sh <nranks>
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
n=${1:-2}
BENCH_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r4/ranks_$n.json 2> gpurun_out/r4/ranks_$n.err || { tail -20 gpurun_out/r4/ranks_$n.err; exit 1; }
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r4/ranks_$n.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['scaling'], round(d['ms_per_step'],4), json.dumps(d['multi_gpu'])[:900])"
