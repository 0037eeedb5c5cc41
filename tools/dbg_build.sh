#!/bin/bash
# rebuild am_lstm_stream.o with extra defines and relink (debug experiments on the GPU box)
set -e
cd "$(dirname "$0")/../nnacousticmodeling_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function --offload-arch=gfx950 $1 -c am_lstm_stream.hip -o am_lstm_stream.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC am_gemm.o am_gemm_split.o am_frontend.o am_lstm_step.o am_lstm_persist.o am_lstm_stream.o am_cells.o am_viterbi.o am_api.o -o ../libamfwd.so
