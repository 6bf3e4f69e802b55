"""Do kernels of two HIP streams run concurrently on this device / driver?  A long bandwidth-bound kernel sequence on a side
stream (normal priority), then a short kernel on the main stream (normal or high priority): how long until the short kernel
has run, compared with the side work's duration."""
import json

import torch


def main():
    dev = torch.device("cuda", 0)
    big = torch.empty(1 << 29, device=dev)      # 2 GiB
    small = torch.empty(1 << 20, device=dev)
    res = {}
    for name, prio in (("main_normal", 0), ("main_high", -1)):
        main_s = torch.cuda.Stream(priority=prio)
        side = torch.cuda.Stream(priority=0)
        for _ in range(2):
            torch.cuda.synchronize()
            e0, e_small, e_side = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(main_s):
                e0.record()
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                for _ in range(4):
                    big.mul_(1.0001)
                e_side.record()
            with torch.cuda.stream(main_s):
                for _ in range(8):
                    small.add_(1.0)
                e_small.record()
            torch.cuda.synchronize()
        res[name] = {"side_work_ms": e0.elapsed_time(e_side), "short_kernels_done_after_ms": e0.elapsed_time(e_small)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
