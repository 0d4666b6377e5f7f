cd $GRAFT_REPO_ROOT
for npt in 2 3 4 5 6 7 8 9; do DCF_LC_KIND=0 DCF_LC_NPT=$npt python3 tools/rw_time.py --std --tag lc0_npt$npt --batch 2 l3 2>&1 | grep -v amdgpu; done
for npt in 3 4 5 6 7 8 10; do DCF_LC_KIND=1 DCF_LC_NPT=$npt python3 tools/rw_time.py --std --tag lc1_npt$npt --batch 2 l3 l4 l5 conv3 2>&1 | grep -v amdgpu; done
for npt in 4 5 6; do DCF_LC_KIND=0 DCF_LC_NPT=$npt python3 tools/rw_time.py --std --tag lc0_npt$npt --batch 8 l3 2>&1 | grep -v amdgpu; done
