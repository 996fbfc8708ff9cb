"""rocprofv3 target: a few training steps of BASELINE config 3 (3x512 BiLSTM, 64 padded
utterances) or its GRU variant.  usage: python3 scripts/prof_bilstm.py [steps] [LSTM|GRU]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cell = sys.argv[2] if len(sys.argv) > 2 else "LSTM"
    print(json.dumps(bench.bilstm_section(torch.device("cuda:0"), 64, steps, cell=cell)))
