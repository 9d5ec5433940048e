cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
echo "old      $(python tools/enc12_ablate.py --only-product 2>&1 | grep product | tail -1)"
echo "ring     $(MI355_ENC12_RING=1 python tools/enc12_ablate.py --only-product 2>&1 | grep product | tail -1)"
echo "ring+c2  $(MI355_ENC12_RING=1 MI355_ENC12_C2=1 python tools/enc12_ablate.py --only-product 2>&1 | grep product | tail -1)"
python tools/enc12_ablate.py 2>&1 | grep "mask 2048\|mask 1024"
done
