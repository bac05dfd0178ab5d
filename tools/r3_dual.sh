for mode in pre der; do echo "k96: $(python3 tools/opbench.py gemmdual 802816 384 96 $mode 20 2>&1 | tail -1)"; done
for mode in pre der; do echo "stage 3: $(python3 tools/opbench.py gemmdual 50176 1536 384 $mode 20 2>&1 | tail -1)"; done
for mode in pre der; do echo "stage 2: $(python3 tools/opbench.py gemmdual 200704 768 192 $mode 20 2>&1 | tail -1)"; done
for mode in pre der; do echo "stage 4: $(python3 tools/opbench.py gemmdual 12544 3072 768 $mode 20 2>&1 | tail -1)"; done
