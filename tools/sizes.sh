#!/bin/bash
python3 tools/prof_chain.py 1024 16384 20 fx16 ca
python3 tools/prof_chain.py 4096 4096 20 fx16 ca
python3 tools/prof_chain.py 256 65536 20 fx16 ca
python3 tools/prof_chain.py 8192 2048 20 fx16 ca
python3 tools/prof_chain.py 4096 4096 20 f32 ca
