O=gpurun_out/r5ab2; mkdir -p $O
A=$1; B=$2
bash tools/ab_ops.sh $A $B 2 > $O/ab_ops.txt 2>&1; grep -E "conv3x3_(fwd|dgrad)|convT" $O/ab_ops.txt | head -60; tail -3 $O/ab_ops.txt
