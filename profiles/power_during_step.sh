#!/bin/bash
# power draw and clocks while the headline training step runs (rocm-smi samples beside `bench.py --steps-only`)
O=gpurun_out/power
mkdir -p $O
rm -f $O/load.txt
rocm-smi --showpower --showclocks --showmaxpower > $O/idle.txt 2>&1
python3 bench.py --steps-only --steps 400 --warmup 5 > $O/bench.json 2> $O/bench.err &
BP=$!
for i in $(seq 1 40); do
  echo "sample $i" >> $O/load.txt
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk\|mclk" >> $O/load.txt
  sleep 0.4
  if ! kill -0 $BP 2>/dev/null; then break; fi
done
wait $BP
cat $O/bench.json
grep -i "Max Graphics" $O/idle.txt
grep -c sample $O/load.txt
grep -i "power (W)" $O/load.txt | awk '{print $NF}' | tr '\n' ' '; echo
grep -i "sclk" $O/load.txt | sed 's/.*(\(.*\))/\1/' | tr '\n' ' '; echo
