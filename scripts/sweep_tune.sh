# launch-plan knobs of the sweeps on the C3 planted factors (scripts/gpu_basis_iter.py prints the per-pair times)
run() { echo "== $*"; env "$@" timeout -k 10 280 python scripts/gpu_basis_iter.py | tail -1; }
run X=1
run IPXK_SWEEP_NARROW=32
run IPXK_SWEEP_NARROW=48
run IPXK_SWEEP_NARROW=160
run IPXK_SWEEP_NARROW=320
run IPXK_SWEEP_XCD_WGS=16
run IPXK_SWEEP_XCD_WGS=64
run IPXK_SWEEP_GRID=512
