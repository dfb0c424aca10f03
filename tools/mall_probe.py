"""Where do `gemm_nt8`'s L2 misses come from -- HBM or the 256 MB Infinity Cache (MALL)?   (run on the GPU box under
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum)

VERDICT r3 item 6 asks for a counter that separates MALL hits from HBM reads.  rocprofv3 -L on gfx950 lists no memory-side
(data-fabric / UMC / MALL) counter at all (blocks: SQ, SPI, TCC, TCP, TA, TD, TCA, CPC, CPF, GRBM, RDC -- the inventory is
kept in profiles/r4_counter_inventory.txt); TCC_EA0_RDREQ_DRAM counts requests routed to local memory as opposed to
xGMI / PCIe, i.e. every miss of a single-GPU run whether the Infinity Cache serves it or not.  What the L2 does expose is
the LATENCY of its misses: TCC_EA0_RDREQ_LEVEL accumulates the number of read requests in flight each cycle, so
LEVEL / RDREQ = mean cycles a miss stays outstanding.  This script produces the two calibration points --
    stream   passes over 8 GiB (every line comes from HBM), on all CUs (HBM saturated: queueing included) and on a
             32-CU masked stream (light load)
    mall     40 passes over 96 MiB (3x the aggregate L2, well inside the 256 MB Infinity Cache: L2 misses, MALL hits),
             likewise on all CUs and on 32
and tools/pmc_latency.py prints mean miss latency per kernel of any PMC run, so the GEMMs of the benchmarked step can be
placed between them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def masked_stream(ncu):
    """stream restricted to the first `ncu` CUs of the mask (tools/cu_partition_bench.py: contiguous mask bits spread evenly
    over the eight XCDs)"""
    import ctypes as C
    hip = C.CDLL('libamdhip64.so')
    words = (C.c_uint32 * 8)()
    for b in range(ncu):
        words[b >> 5] |= 1 << (b & 31)
    s = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)


def main():
    dev = 'cuda'
    big = torch.empty(2 << 30, device=dev, dtype=torch.float32)      # 8 GiB
    big.fill_(1.0)
    small = torch.empty(24 << 20, device=dev, dtype=torch.float32)   # 96 MiB
    small.fill_(1.0)
    torch.cuda.synchronize()
    # four reductions = four kernel names (the reduction functor is a template argument); each reads its input once per call
    #   sum   8 GiB, all CUs     HBM at full load          amin  8 GiB, 32 CUs     HBM, lightly loaded
    #   amax  96 MiB x 40, all   Infinity Cache, loaded    prod  96 MiB x 40, 32   Infinity Cache, lightly loaded
    for _ in range(3):
        big.sum()
    for _ in range(40):
        small.amax()
    few = masked_stream(32)
    few.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(few):
        big.amin()
        for _ in range(40):
            small.prod()
    torch.cuda.synchronize()
    print('done')


if __name__ == '__main__':
    main()
