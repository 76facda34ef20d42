enc() { echo $(( $1 | ($2 << 8) | ($3 << 16) | ($4 << 24) )); }
for cfg in "0 0 12 24" "0 24 12 24" "4 14 12 24" "6 12 12 24" "6 12 8 24" "4 16 16 24" "6 12 12 20" "8 10 14 20" "4 20 10 32"; do
  set -- $cfg
  v=$(enc $1 $2 $3 $4)
  echo -n "from $1 count $2 len $3 main $4: "
  for p in 1270 4096 254; do ECAL_ADAPTIVE_SHAPE=side=$v python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -1 | awk '{printf "%s pieces %s s | ", $6, $8}'; done; echo
done
