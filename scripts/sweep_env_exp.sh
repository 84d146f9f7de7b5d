# launch-plan knobs of the sweeps at C3 (planted factors): per-iteration time and the two pairs
cd $GRAFT_REPO_ROOT
for setting in "" "IPXK_SWEEP_GRID=128" "IPXK_SWEEP_GRID=192" "IPXK_SWEEP_NARROW=192" "IPXK_SWEEP_NARROW=384" "IPXK_SWEEP_NARROW=192 IPXK_SWEEP_XCD_WGS=64" "IPXK_SWEEP_NARROW=48" "IPXK_SWEEP_GRID=192 IPXK_SWEEP_NARROW=192"; do
  echo "== $setting"
  env $setting python scripts/gpu_basis_iter.py 2>&1 | grep "profiled"
done
