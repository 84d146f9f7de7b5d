cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-banded --no-basis --no-newton --no-other-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_trace -- $B > gpurun_out/r02_trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_fetch -- $B > gpurun_out/r02_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_write -- $B > gpurun_out/r02_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_basis_trace -- python3 scripts/gpu_basis_iter.py > gpurun_out/r02_basis_trace.log 2>&1 &&
python3 scripts/trace_summary.py gpurun_out/r02_basis_trace > gpurun_out/r02_basis_summary.txt &&
python3 scripts/trace_iteration.py gpurun_out/r02_basis_trace > gpurun_out/r02_basis_iteration.txt &&
tail -3 gpurun_out/r02_basis_trace.log && tail -25 gpurun_out/r02_basis_iteration.txt
