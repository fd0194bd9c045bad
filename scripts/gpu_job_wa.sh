timeout 300 python -m pytest tests/test_word_timing.py tests/test_dtw.py -q -m gpu -x 2>&1 | tail -15
