"""Where do `gemm_nt8`'s L2 misses come from -- HBM or the 256 MB Infinity Cache (MALL)?   (run on the GPU box under
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum)

VERDICT r3 item 6 asks for a counter that separates MALL hits from HBM reads.  rocprofv3 -L on gfx950 lists no memory-side
(data-fabric / UMC / MALL) counter at all (blocks: SQ, SPI, TCC, TCP, TA, TD, TCA, CPC, CPF, GRBM, RDC -- the inventory is
kept in profiles/r4_counter_inventory.txt); TCC_EA0_RDREQ_DRAM counts requests routed to local memory as opposed to
xGMI / PCIe, i.e. every miss of a single-GPU run whether the Infinity Cache serves it or not.  What the L2 does expose is
the LATENCY of its misses: TCC_EA0_RDREQ_LEVEL accumulates the number of read requests in flight each cycle, so
LEVEL / RDREQ = mean cycles a miss stays outstanding.  This script produces the two calibration points --
    stream   one pass over 8 GiB (every line comes from HBM)
    mall     40 passes over 96 MiB (3x the aggregate L2, well inside the 256 MB Infinity Cache: L2 misses, MALL hits)
and tools/pmc_latency.py prints mean miss latency per kernel of any PMC run, so the GEMMs of the benchmarked step can be
placed between them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    dev = 'cuda'
    big = torch.empty(2 << 30, device=dev, dtype=torch.float32)      # 8 GiB
    big.fill_(1.0)
    small = torch.empty(24 << 20, device=dev, dtype=torch.float32)   # 96 MiB
    small.fill_(1.0)
    torch.cuda.synchronize()
    # reductions read their input exactly once per call: the kernel name carries the size through the launch count
    s = 0.0
    for _ in range(3):
        s += float(big.sum())           # 'stream': reduce_kernel over 8 GiB
    for _ in range(40):
        small.sum()                     # 'mall' (after the first pass): reduce_kernel over 96 MiB
    torch.cuda.synchronize()
    print('done', s)


if __name__ == '__main__':
    main()
