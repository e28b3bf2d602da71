cp mmhand_amd/libmmhand_hip.so /tmp/new.so
for r in 1 2; do
for v in old new; do
  if [ $v = old ]; then cp mmhand_amd/libmmhand_hip_old.so mmhand_amd/libmmhand_hip.so; else cp /tmp/new.so mmhand_amd/libmmhand_hip.so; fi
  echo "== $v"; python tools/ab_step.py opt:lp16_persist 1 1 2>&1 | grep "ms/step"
done; done
