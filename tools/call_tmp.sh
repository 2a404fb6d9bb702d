python tools/sq_fractions.py r4h frames 2>&1 | tail -1
python tools/sq_fractions.py r4h frames 2>&1 | tail -1
