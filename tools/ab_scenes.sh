# A/B of the same variants over the three reference scenes and the 1 M-sphere plane (development aid)
set -e
for s in scene1 scene2 scene3; do
  echo "== $s"
  AB_SCENE=fixtures/$s.ron AB_F=${AB_F:-256} timeout -k 10 300 python -u tools/ab.py "$@" | grep -v parity
done
echo "== 1M spheres 4K"
AB_SPHERES=1000000 AB_W=3840 AB_H=2160 AB_F=16 timeout -k 10 300 python -u tools/ab.py "$@" | grep -v parity
