set +e
R=$PWD
for L in prev new prev new; do
  if [ $L = new ]; then unset RRL_HIP_LIB; else export RRL_HIP_LIB=$R/profiles/_ab_$L.so; fi
  python profiles/stage_times.py $L 2>/dev/null | tail -1
done
