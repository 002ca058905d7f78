for f in 1024 2048 4096 8192 16384; do python3 tools/prof_chain.py 4096 $f 20; done
