#!/usr/bin/env python3
"""What a fork / join between two branches of ONE hipGraph costs on this stack, with spin kernels of known length (one block
each: the branches do not compete for anything).  Per repetition: head (A us) -> two branches of B us each -> join.

    python scripts/probe_graph_fork.py
"""
import time

import torch


def main():
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    s1, s2, s3 = (torch.cuda.Stream(device=device) for _ in range(3))
    # calibrate the spin kernel: cycles per microsecond
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    torch.cuda._sleep(10_000_000)
    b.record()
    torch.cuda.synchronize()
    per_us = 10_000_000 / (a.elapsed_time(b) * 1e3)
    print(f"spin kernel: {per_us:.1f} cycles per us")

    def spin(us):
        torch.cuda._sleep(int(us * per_us))

    def capture(body, reps=20):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s1):
            for _ in range(reps):
                body()
        return graph

    def time_graph(graph, reps=20, replays=10):
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(replays):
            graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / replays / reps * 1e6

    def serial(head, branch):
        def body():
            spin(head)
            spin(branch)
            spin(branch)
        return body

    def forked(head, branch, second=None, tail=0.0):
        second = branch if second is None else second

        def body():
            spin(head)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                spin(second)
            spin(branch)
            s1.wait_stream(s2)
            if tail:
                spin(tail)
        return body

    def forked_many_kernels(head, branch, pieces):
        """each branch as `pieces` dependent kernels (like a chain of GEMMs)"""
        def body():
            spin(head)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                for _ in range(pieces):
                    spin(branch / pieces)
            for _ in range(pieces):
                spin(branch / pieces)
            s1.wait_stream(s2)
        return body

    def both_forked(head, branch):
        """head on s1, BOTH branches on side streams (s2, s3), join on s1"""
        def body():
            spin(head)
            s2.wait_stream(s1)
            s3.wait_stream(s1)
            with torch.cuda.stream(s2):
                spin(branch)
            with torch.cuda.stream(s3):
                spin(branch)
            s1.wait_stream(s2)
            s1.wait_stream(s3)
        return body

    def leapfrog(branch, piece):
        """no join: each stream runs its branch, a short piece (its half of the gradient assembly), waits for the OTHER stream's
        piece (an event edge), runs a second short piece (its half of the optimizer step) and goes on with its next branch"""
        state = {}

        def body():
            first = "e1" not in state
            if first:
                s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                spin(branch)
                spin(piece)
                e2 = torch.cuda.Event()
                e2.record(s2)
            spin(branch)
            spin(piece)
            e1 = torch.cuda.Event()
            e1.record(s1)
            s1.wait_event(e2)
            spin(piece)
            with torch.cuda.stream(s2):
                s2.wait_event(e1)
                spin(piece)
            state["e1"] = e1
        return body, state

    def capture_leapfrog(branch, piece, reps=20):
        body, state = leapfrog(branch, piece)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s1):
            for _ in range(reps):
                body()
            s1.wait_stream(s2)
        return graph

    for branch in (100, 30):
        print(f"leapfrog (no join; two event edges per repetition), branches {branch} us + 2 x 5 us pieces per stream: "
              f"{time_graph(capture_leapfrog(branch, 5)):8.1f} us per repetition (ideal {branch + 10})")
    for head, branch in ((15, 100), (15, 30), (5, 100)):
        print(f"head {head} us, branches {branch} us each:")
        print(f"  serial (head + 2 branches on one stream)      {time_graph(capture(serial(head, branch))):8.1f} us per repetition (ideal {head + 2 * branch})")
        print(f"  fork / join                                    {time_graph(capture(forked(head, branch))):8.1f} us (ideal {head + branch})")
        print(f"  fork / join, forked branch 20 us shorter       {time_graph(capture(forked(head, branch, branch - 20))):8.1f} us (ideal {head + branch})")
        print(f"  fork / join, main branch 20 us shorter         {time_graph(capture(forked(head, branch - 20, branch))):8.1f} us (ideal {head + branch})")
        print(f"  fork / join, 9 kernels per branch              {time_graph(capture(forked_many_kernels(head, branch, 9))):8.1f} us (ideal {head + branch} + 8 boundaries)")
        print(f"  both branches forked                           {time_graph(capture(both_forked(head, branch))):8.1f} us (ideal {head + branch})")
    # the same fork / join issued eagerly (no graph), for reference
    body = forked(15, 100)
    with torch.cuda.stream(s1):
        for _ in range(5):
            body()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            body()
        torch.cuda.synchronize()
    print(f"eager fork / join (15 + 100): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per repetition")


if __name__ == "__main__":
    main()
