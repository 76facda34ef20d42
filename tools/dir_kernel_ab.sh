for rep in 1 2 3; do
for g in 0 1; do
  echo -n "dir kernel $g: "
  for p in 1270 4096; do
    if [ $g = 1 ]; then export ECAL_ADAPTIVE_DIR_KERNEL=1; else unset ECAL_ADAPTIVE_DIR_KERNEL; fi
    python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -2 | tr '\n' ' ' | awk '{printf "%s %s pieces %s s | ", $3, $9, $11}'; done; echo
done; done
