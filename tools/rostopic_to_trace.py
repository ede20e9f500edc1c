"""rostopic dumps -> trace file.

    rostopic echo -b run.bag -p /odom                 > odom.csv
    rostopic echo -b run.bag -p /out/landmarks/sensor > landmarks.csv
    python tools/rostopic_to_trace.py odom.csv landmarks.csv run.asltrc [--freq 1.0] [--t-start-ns N]

The result is what `aslam_trace_file_open` (include/aslam_trace_file.h) reads and `bench.py` / `aslam_replay` consume."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from awesomeslam_amd import rosdump

ap = argparse.ArgumentParser()
ap.add_argument("odom_csv"); ap.add_argument("landmarks_csv"); ap.add_argument("out")
ap.add_argument("--freq", type=float, default=1.0, help="the node's spin rate (config.h FREQ)")
ap.add_argument("--t-start-ns", type=int, default=None, help="time of the spin before the first one (default: first message)")
a = ap.parse_args()
tr = rosdump.to_trace(open(a.odom_csv).read(), open(a.landmarks_csv).read(), a.freq, a.t_start_ns)
tr.to_file(a.out, with_truth=False)
print(f"{a.out}: {tr.T} callbacks, {int(tr.obs_new.sum())} sensor messages, max {tr.max_obs} observations per message")
