#!/bin/bash
set -e
hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bench_igemm.cpp -o /tmp/bi_0 2>/dev/null
/tmp/bi_0
