cd $GRAFT_REPO_ROOT
for v in "" ng8 gb50 ng2; do
  if [ -n "$v" ]; then export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so.$v; else unset NELE_LIB; fi
  echo "variant=$v"; bash tools/prof_one.sh tools/siib_ab.py 256 63871 2>&1 | grep -E "siib_spec"
done
