for i in 1 2 3; do python tools/debug_diff.py 1920 1080 64 5 2>&1 | grep "^picture" | tr '\n' ' '; echo; done
