# shader clock / power while a kernel loop runs: tools/clk.sh <opbench args...>
python3 tools/opbench.py "$@" > /dev/null 2>&1 &
pid=$!
sleep ${CLK_DELAY:-6}
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -6
wait $pid
