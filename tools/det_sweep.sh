for i in 1 2 3 4 5 6 7 8 9 10 11 12; do CVD_DEBUG_COARSE=1 DET_BENCH=1 python tools/det_check.py 6 2>&1 | grep "^20\|^3 \|fail 1" | cut -c1-150 | tr "\n" "|"; echo; done
