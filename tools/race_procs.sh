cd $GRAFT_REPO_ROOT
n=${1:-60}
for wg in 1 0; do
  bad=0
  for i in $(seq 1 $n); do
    out=$(TESTBATCH=1 WG=$wg python tools/race_hunt.py 8 2>&1 | grep -v amdgpu.ids)
    if echo "$out" | grep -q "overflow=\|padding"; then bad=$((bad+1)); echo "wg=$wg run $i:"; echo "$out" | head -4 | cut -c1-500; fi
  done
  echo "wg=$wg: $bad bad processes of $n"
done
