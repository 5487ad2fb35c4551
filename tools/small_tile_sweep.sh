for thr in 0 1000; do for n in 53 110 160 220 300 480; do
echo -n "threshold $thr: "; DPTM_SMALL_TILE_SITES=$thr timeout -k 10 100 python tools/small_call_timeline.py $n 2>/dev/null | grep "back to back" | sed "s/^/$n sites /"
done; done
