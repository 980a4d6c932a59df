# LDS node-record stride A/B (80 = the record itself; more = padding): one process per library build.
for nb in 80 88 96 104 112; do
  lib=pathtracer_amd/libmi355pt_n$nb.so; [ $nb = 80 ] && lib=pathtracer_amd/libmi355pt.so
  echo "stride $nb"
  MI355PT_LIB=$lib AB_F=256 AB_ROUNDS=5 timeout -k 10 120 python tools/ab.py "s$nb=MI355PT_BLOCK=1024" 2>&1 | tail -1 || exit 1
done
