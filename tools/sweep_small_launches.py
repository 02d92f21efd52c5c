"""Where does recon_wide_kernel (four waves per chunk) stop paying?  Launches of k 1080p pictures (k streams, a picture per
launch) timed through bench.py's leg (HIP events over 100 launches, 40 ms of the same launches in front for the clocks) with
whatever libmpeghip.so is in place: the caller (tools/ab/small_launch_sweep.sh) swaps in builds whose launch_batch never /
always takes the wide kernel.   usage: python tools/sweep_small_launches.py <label> [k ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    label = sys.argv[1]
    ks = [int(k) for k in sys.argv[2:]] or [1, 2, 3, 4, 6, 8, 16]
    sys.argv = [sys.argv[0], "--check", "0"]
    import bench
    import torch
    from mpeg_amd import abi

    args = bench.parse_args()
    torch.cuda.set_device(0)
    tstream = torch.cuda.Stream(device=0)
    ctx = abi.Context(0, tstream.cuda_stream)
    for rgba in [bool(int(x)) for x in os.environ.get("SWEEP_RGBA", "1,0").split(",")]:
        for profile in os.environ.get("SWEEP_PROFILES", "typical,dense").split(","):
            row = []
            for k in ks:
                leg = bench.video_leg(ctx, args, profile, rgba, k, steps=100, ramp_ms=40.0)
                row.append("%d: %.2f" % (k, leg["roofline"]["avg_launch_ms"] * 1e3))
            print("%-8s %-7s %-5s us per launch of k pictures  %s" % (label, profile, "rgba" if rgba else "plain", "  ".join(row)), flush=True)


if __name__ == "__main__":
    main()
