python tools/classify_stats.py
