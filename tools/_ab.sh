L=gpurun_out/ab15.log
: > $L
for i in 1 2 3; do
  for pos in 0 1 2 3; do
  echo "f32 pos $pos" >> $L; DV_TMP_BNPOS=$pos python tools/bf16_bench.py 256 300 0 2>/dev/null >> $L
  echo "bf16 pos $pos" >> $L; DV_TMP_BNPOS=$pos python tools/bf16_bench.py 256 300 2>/dev/null >> $L
  done
done
