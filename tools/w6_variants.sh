#!/bin/bash
# GPU box: time the BiLSTM forward recurrences for each library variant named on the command line ("base" = the product library,
# "old" = the product library with RLT_LSTM6W=0)
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  echo "== $V"
  if [ "$V" = base ]; then LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/librlt_hip.so; E=1;
  elif [ "$V" = old ]; then LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/librlt_hip.so; E=0;
  else LIB=$GRAFT_REPO_ROOT/ranked-list-truncation_amd/csrc/variants/librlt_$V.so; E=1; fi
  RLT_LSTM6W=$E RLT_HIP_LIB=$LIB timeout -k 10 120 python3 tools/bench_kernels.py ${W6_BENCH:-lstm_w} 2>&1 | grep bilstm || exit 1
done
