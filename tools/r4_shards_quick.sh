mkdir -p gpurun_out/r4z
{
for w in 1 2 4 8; do echo "world $w"; EXP_FIRST_X_ONLY=1 HXR_TILE=$([ $w = 1 ] && echo 64 || echo 8) timeout -k 10 300 python3 tools/exp_shard_sched.py $w 2>&1 | grep "slowest" | head -1; done
for wl in smallpt16 heightfield16 cornell16; do echo $wl; HXR_TILE=8 HXR_ALL_RANKS=1 timeout -k 10 300 python3 tools/shard_time.py $wl 2>&1 | grep "world 8" ; HXR_SCHED_FIRST_X=6 HXR_TILE=8 HXR_ALL_RANKS=1 timeout -k 10 300 python3 tools/shard_time.py $wl 2>&1 | grep "world 8" | sed "s/^/first_x 6: /"; done
} > gpurun_out/r4z/retuned.txt 2>&1
cat gpurun_out/r4z/retuned.txt
