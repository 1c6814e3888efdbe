cd $GRAFT_REPO_ROOT
cp hero_amd/libhero_hip.so /tmp/base.so
for v in base rpw2 rpw4; do
  if [ $v != base ]; then cp hero_amd/libhero_$v.so hero_amd/libhero_hip.so; else cp /tmp/base.so hero_amd/libhero_hip.so; fi
  echo "== $v"; timeout 120 python tools/ln_bench.py 2>&1 | grep rows
done
