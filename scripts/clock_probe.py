"""Clock levels (sysfs pp_dpm_*) sampled every ~1 ms while BiLSTM training steps run: do the persistent
recurrences (latency-bound, little memory traffic) let a clock domain fall that the GEMMs behind them
then wait for?  usage (GPU box): python3 scripts/clock_probe.py"""
import collections
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
props = torch.cuda.get_device_properties(0)
addr = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
card = [c for c in glob.glob("/sys/class/drm/card*/device") if os.path.realpath(c).endswith(addr)][0]
files = [f for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk") if os.path.isfile(card + "/" + f)]
print("card", card, "files", files)
for f in files:
    print(f, open(card + "/" + f).read().replace("\n", " | "))
stop = False
hist = {f: collections.Counter() for f in files}


def loop():
    while not stop:
        for f in files:
            try:
                for line in open(card + "/" + f):
                    if "*" in line:
                        hist[f][line.strip()] += 1
            except OSError:
                pass
        time.sleep(0.001)


th = threading.Thread(target=loop, daemon=True)
th.start()
bench.bilstm_section(dev, 64, 6, cell="LSTM")
stop = True
th.join()
for f in files:
    print(f, dict(hist[f]))
