# round-5 profiles: run on the GPU box from the repo root (gpurun -- 'bash scripts/prof_r05.sh')
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-banded --no-basis --no-newton --no-other-configs --no-lu --no-maxvolume --no-dropin"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_trace -- $B > gpurun_out/r05_trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r05_fetch -- $B > gpurun_out/r05_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r05_write -- $B > gpurun_out/r05_write.log 2>&1 &&
python3 scripts/make_profile_summary.py gpurun_out/r05_trace gpurun_out/r05_fetch gpurun_out/r05_write r05 &&
cp profiles/r05_* profiles/pmc_traffic.json gpurun_out/ &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_basis_trace -- python3 scripts/gpu_basis_iter.py > gpurun_out/r05_basis_trace.log 2>&1 &&
python3 scripts/trace_summary.py gpurun_out/r05_basis_trace > gpurun_out/r05_basis_kernel_summary.txt &&
python3 scripts/trace_iteration.py gpurun_out/r05_basis_trace > gpurun_out/r05_basis_iteration.txt &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r05_basis_fetch -- python3 scripts/gpu_basis_iter.py > gpurun_out/r05_basis_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r05_basis_write -- python3 scripts/gpu_basis_iter.py > gpurun_out/r05_basis_write.log 2>&1 &&
python3 scripts/pmc_iteration.py gpurun_out/r05_basis_fetch gpurun_out/r05_basis_write acc acc > gpurun_out/r05_basis_pmc_traffic.txt &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_c2_trace -- python3 scripts/gpu_probe_c2.py > gpurun_out/r05_c2_trace.log 2>&1 &&
python3 scripts/trace_c2_iteration.py gpurun_out/r05_c2_trace > gpurun_out/r05_c2_iteration.txt &&
python3 scripts/trace_summary.py gpurun_out/r05_c2_trace > gpurun_out/r05_c2_kernel_summary.txt &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_lu_trace -- python3 scripts/gpu_lu_bench.py 100000 220000 8000 > gpurun_out/r05_lu_trace.log 2>&1 &&
python3 scripts/trace_summary.py gpurun_out/r05_lu_trace > gpurun_out/r05_lu_bump8000_kernel_summary.txt &&
tail -3 gpurun_out/r05_basis_trace.log && tail -12 gpurun_out/r05_c2_iteration.txt && tail -4 gpurun_out/r05_basis_pmc_traffic.txt && head -12 gpurun_out/r05_lu_bump8000_kernel_summary.txt
