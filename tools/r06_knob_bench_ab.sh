python tools/calibrate.py 2>&1 | tail -1
python bench.py --steps 5 --warmup 2 --no-extras > /dev/null 2>&1
for rep in 1 2; do
for v in "INSMOS_CONV_ROW32=1" "INSMOS_CONV_ROW32=0" "INSMOS_CONV_ROW32=2" "INSMOS_CONV_ROWLANE=0" "INSMOS_CONV_ROWLANE=3"; do
  r=$(env $v timeout 300 python bench.py --timed-only --steps 20 --warmup 3 2>/dev/null | grep -o '"value": [0-9.]*')
  echo "$v: $r"
done; done
