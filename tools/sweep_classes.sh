ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in default c1 c4 c16; do
  for x in 16 32 64; do
    if [ $v = default ]; then unset GM_LIB_PATH; else export GM_LIB_PATH=$ROOT/build/variants/libgm_hip_$v.so; fi
    export GM_NORMALS_XCD=$x
    k=$(python tools/stage_times.py --reps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stage_ms']['normals'])")
    b=$(python bench.py --steps 100 --warmup 5 --no-secondary --no-cpu-baseline --group-points 0 --fixed-slots 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4))")
    echo "classes=$v xcd=$x normals_ms=$k step_ms=$b"
  done
done
