for sx in 8 16 32 64; do for fx in 2 4 8 16; do
  echo -n "split $sx first $fx: "; HXR_SCHED_SPLIT_X=$sx HXR_SCHED_FIRST_X=$fx python3 tools/run_frame.py meshes16 6 2>&1 | grep "kernel ms" | awk '{print $(NF-2), $(NF-1), $NF}'
done; done
