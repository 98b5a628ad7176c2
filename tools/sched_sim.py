"""Scheduling simulator for the per-instance kernels: greedy dispatch of workgroups in grid order onto S slots (the hardware
dispatcher), iteration counts from gpurun_out/iters_*.npy (tools/iter_hist.py).  Compares grid order, longest-first, and
turn-based continuation (every instance runs at most T iterations per launch, unfinished ones continue in the next)."""
import heapq, sys
import numpy as np

def launch(durs, S, t0=0.0):
    """durs in dispatch order -> finish time of the launch (greedy: next workgroup to the first free slot)."""
    free = [t0] * S
    heapq.heapify(free)
    end = t0
    for d in durs:
        s = heapq.heappop(free)
        e = s + d
        end = max(end, e)
        heapq.heappush(free, e)
    return end

def turns(it, S, t_it, T, ovh, gap, exit_cost=0.3):
    """every launch: all B workgroups; finished instances cost exit_cost us; others run min(T, remaining) iterations (+ovh reload)."""
    rem = it.astype(float).copy()
    t = 0.0
    nl = 0
    while (rem > 0).any():
        Tk = T[min(nl, len(T) - 1)]
        d = np.where(rem > 0, np.minimum(rem, Tk) * t_it + ovh, exit_cost)
        t = launch(d, S, t) + gap
        rem = np.maximum(rem - Tk, 0)
        nl += 1
    return t - gap, nl

if __name__ == "__main__":
    path, S, t_it = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])      # t_it in us per iteration
    it = np.load(path)
    ovh = float(sys.argv[4]) if len(sys.argv) > 4 else 6.0
    print("B=%d mean %.1f max %d ; ideal %.3f ms" % (len(it), it.mean(), it.max(), it.sum() * t_it / S / 1e3))
    print("grid order   %.3f ms" % (launch(it * t_it + ovh, S) / 1e3))
    print("longest-first %.3f ms" % (launch(np.sort(it)[::-1] * t_it + ovh, S) / 1e3))
    for T in ([50], [75], [100], [125], [150], [175], [200], [100, 50], [125, 50], [150, 50], [150, 25], [125, 25], [100, 25],[175,25]):
        e, nl = turns(it, S, t_it, T, ovh, 5.0)
        print("turns %-10s %.3f ms (%d launches)" % (T, e / 1e3, nl))

def launch_xcd(durs, S, nx=8, t0=0.0):
    """workgroup i goes to XCD i % nx (static round-robin), each XCD dispatches greedily onto its S / nx slots"""
    return max(launch(durs[x::nx], S // nx, t0) for x in range(nx))
