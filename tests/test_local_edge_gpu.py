"""The in-process device hop (`ops/csrc/comm/local_edge.cpp`): what multi-stage pipelines inside one
process run over on the GPU, and what the native stage loop is exercised with. Checked against plain
tensor copies: byte-exact delivery in order, ordering against a busy producer stream, more messages
than staging slots with a receiver that lags, refusal of a message of the wrong size, the receive
deadline."""
import threading
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def make(max_bytes=1 << 20, timeout_s=20.0):
    from distributed_llm_inference_amd.parallel.local_edge import LocalDeviceEdge
    return LocalDeviceEdge("cuda:0", max_bytes, timeout_s=timeout_s)


def test_round_trip_is_byte_exact_and_ordered_behind_the_producer():
    e = make(1024 * 4096 * 2)
    try:
        big = torch.randn(4096, 4096, device="cuda")
        src = torch.zeros(1024, 4096, device="cuda", dtype=torch.bfloat16)
        dst = torch.zeros_like(src)
        want = []
        for i in range(3):
            (big @ big)                                   # keeps the producer stream busy ahead of the write below
            src.fill_(float(i + 1))
            want.append(float(i + 1))
            hs = e.send(src)
            hs.wait()                                     # the next fill_ must not overtake the copy out of src
            hr = e.recv(dst)
            hr.wait()
            assert float(dst.float().mean().item()) == want[-1]
        ids = torch.arange(1024, device="cuda", dtype=torch.int32)
        out = torch.zeros_like(ids)
        e.send(ids).wait()
        h = e.recv(out)
        h.wait()
        torch.cuda.synchronize()
        assert torch.equal(out, ids) and h.done()
    finally:
        e.close()


def test_more_messages_than_staging_slots_with_a_lagging_receiver():
    e = make(64 * 1024)
    n = 40                                                # 5 times round the staging ring
    got = []

    def receiver():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(n):
                if i % 7 == 0:
                    time.sleep(0.01)                      # lag: the sender has to wait for free slots
                buf = torch.zeros(16 * 1024, device="cuda", dtype=torch.int32)
                e.recv(buf).wait()
                got.append(buf)
            s.synchronize()

    th = threading.Thread(target=receiver)
    th.start()
    try:
        sends = []                                        # a handle keeps its source alive until the copy out of it has run
        for i in range(n):
            sends.append(e.send(torch.full((16 * 1024,), i, device="cuda", dtype=torch.int32)))
        th.join(60)
        assert not th.is_alive()
        torch.cuda.synchronize()
        assert [int(b[0].item()) for b in got] == list(range(n)) and all(int(b.min()) == int(b.max()) for b in got)
    finally:
        e.close()


def test_wrong_size_and_deadline_are_errors_not_hangs():
    from distributed_llm_inference_amd.parallel.local_edge import LocalEdgeError
    e = make(4096, timeout_s=0.3)
    try:
        with pytest.raises(LocalEdgeError, match="exceeds"):
            e.send(torch.zeros(8192, device="cuda", dtype=torch.uint8))
        e.send(torch.zeros(64, device="cuda", dtype=torch.uint8))
        with pytest.raises(LocalEdgeError, match="expected"):
            e.recv(torch.zeros(32, device="cuda", dtype=torch.uint8))
        t0 = time.perf_counter()
        e2 = make(4096, timeout_s=0.3)
        with pytest.raises(LocalEdgeError, match="no message"):
            e2.recv(torch.zeros(32, device="cuda", dtype=torch.uint8))
        assert time.perf_counter() - t0 < 5.0
        e2.abort()
        with pytest.raises(LocalEdgeError, match="aborted"):
            e2.send(torch.zeros(32, device="cuda", dtype=torch.uint8))
        e2.close()
    finally:
        e.close()
