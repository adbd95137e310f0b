cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_xcd16_probe5.log; : > $O
for r in 15 12 8; do
echo "=== BX3=1 RPX=$r" >> $O
BX3=1 RPX=$r HID=512 timeout 200 tools/xcd_chain_bench.bin >> $O 2>&1; echo "rc $?" >> $O
done
grep "===\|vs CPU\|1 launch\|rc " $O | grep -v "column-split" | cut -c1-230
