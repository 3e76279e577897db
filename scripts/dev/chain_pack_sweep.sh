cd $GRAFT_REPO_ROOT
export PISA_HIP_LIB=${PISA_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/pisa_amd/libpisa_hip_dev.so}   # development build: make -C pisa_amd/csrc dev
for cfg in "14 0" "10 0" "8 0" "18 0" "12 4" "14 6" "10 6" "24 0" "14 0"; do
  set -- $cfg
  export PISA_HIP_PACK_T4=$1 PISA_HIP_PACK_T2=$2
  echo "T4 $1 T2 $2: $(bash scripts/dev/step_timeline.sh --steps 300 2>/dev/null | grep -E "^chain|^step|^terms" | awk '{printf "%s %s  ", $1, $3}')"
done
