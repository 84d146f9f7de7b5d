# round-3 profiles: run on the GPU box from the repo root (gpurun -- 'bash scripts/prof_r03.sh')
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-banded --no-basis --no-newton --no-other-configs --no-lu --no-maxvolume"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_trace -- $B > gpurun_out/r03_trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03_fetch -- $B > gpurun_out/r03_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03_write -- $B > gpurun_out/r03_write.log 2>&1 &&
python3 scripts/make_profile_summary.py gpurun_out/r03_trace gpurun_out/r03_fetch gpurun_out/r03_write r03 &&
cp profiles/r03_* profiles/pmc_traffic.json gpurun_out/ &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_basis_trace -- python3 scripts/gpu_basis_iter.py > gpurun_out/r03_basis_trace.log 2>&1 &&
python3 scripts/trace_summary.py gpurun_out/r03_basis_trace > gpurun_out/r03_basis_kernel_summary.txt &&
python3 scripts/trace_iteration.py gpurun_out/r03_basis_trace > gpurun_out/r03_basis_iteration.txt &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03_basis_fetch -- python3 scripts/gpu_basis_iter.py > gpurun_out/r03_basis_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03_basis_write -- python3 scripts/gpu_basis_iter.py > gpurun_out/r03_basis_write.log 2>&1 &&
python3 scripts/pmc_iteration.py gpurun_out/r03_basis_fetch gpurun_out/r03_basis_write > gpurun_out/r03_basis_pmc_traffic.txt &&
tail -3 gpurun_out/r03_basis_trace.log && tail -25 gpurun_out/r03_basis_iteration.txt && tail -4 gpurun_out/r03_basis_pmc_traffic.txt
