# rocprofv3 kernel summary of the reference's LpSolver through the Hip classes on a synthetic LP (run on the GPU box from the repo root):
#   bash scripts/gpu_lp_dropin_profile.sh m n
M=${1:-12000}; N=${2:-30000}; TAG=${3:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=$(python3 - <<PY
import sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_lp_dropin as T
d = tempfile.mkdtemp()
T.write_model(d + "/in", *T.general_lp($M, $N, 31), crossover=0)
print(d)
PY
)
mkdir -p $D/out
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_lp_$M -o lp -- oracle/_ref/test_lp_hip $D/in $D/out > gpurun_out/${TAG}_lp_profile_${M}.log 2>&1
python3 scripts/rocprof_top.py $(find /tmp/prof_lp_$M -name "*.db" | head -1) 30 > gpurun_out/${TAG}_lp_profile_${M}_kernel_summary.txt
head -1 gpurun_out/${TAG}_lp_profile_${M}.log | cut -c1-300
cat gpurun_out/${TAG}_lp_profile_${M}_kernel_summary.txt
