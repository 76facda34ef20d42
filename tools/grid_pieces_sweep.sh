for rep in 1 2 3; do
for g in 256 2000; do
  echo -n "grid pieces $g: "
  for p in 1270 4096; do ECAL_ADAPTIVE_GRID_PIECES=$g python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -1 | awk '{printf "%s pieces %s s %s passes | ", $6, $8, $10}'; done; echo
done; done
