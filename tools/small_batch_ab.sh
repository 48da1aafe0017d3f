O=gpurun_out/small; mkdir -p $O
for b in 8 16 32 64 128; do
  st=$(( 12800 / b )); [ $st -gt 400 ] && st=400
  echo "## mobilenet1.0 batch $b"
  python tools/ab.py --rounds 3 --args "--batch-size $b --steps $st" "before (round-6 tree without the two rules)|FQ_LIB_PATH=vtmp/lib_before_small.so" "final" 2>&1 | tail -3
done > $O/ab.txt 2>&1
for b in 16 32; do
  echo "## resnet50_v1 channel online batch $b"
  python tools/ab.py --rounds 2 --args "--model resnet50_v1 --quant-type channel --batch-size $b --steps 150" "before|FQ_LIB_PATH=vtmp/lib_before_small.so" "final" 2>&1 | tail -3
done >> $O/ab.txt 2>&1
cat $O/ab.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
