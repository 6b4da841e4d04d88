"""Body sharding across ranks (one process per GPU) -- a Python MODEL of the host logic, TEST INFRASTRUCTURE only: the product's
arithmetic is C++ (csrc/mapn_context.cpp, mapn_sym_plan.cpp, mapn_shard.cpp); the tests hold it against this restatement.

Rank p of P owns the contiguous slice [p*N/P, (p+1)*N/P) of positions and velocities; every
rank keeps a full replica of the position buffer and the new float4 slices are all-gathered
once per step.  The reference has no data sharding -- its two adapters split
compute from rendering (Particles.cpp:130-133, 212-243) -- so this is the host logic the
multi-GPU step adds; the same arithmetic lives in csrc/mapn_context.cpp (create_common,
enqueue_step) and is cross-checked against it in tests.
"""
from __future__ import annotations

from dataclasses import dataclass

BLOCK_SIZE = 64   # defines.h:37


def active_bodies(num_active: int, n: int) -> int:
    """Compute.cpp:1041: Dispatch(ceil(numActive/64)) groups of 64 threads, clipped to N."""
    if num_active <= 0:
        return 0
    return min((num_active + BLOCK_SIZE - 1) // BLOCK_SIZE * BLOCK_SIZE, n)


def shard_range(n: int, rank: int, world_size: int) -> tuple[int, int]:
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} / world_size {world_size}")
    if n % world_size:
        raise ValueError(f"world_size {world_size} must divide num_particles {n}")
    count = n // world_size
    return rank * count, count


def remote_segments(n: int, rank: int, world_size: int) -> list[tuple[int, int]]:
    """j-ranges a rank needs from the other ranks: before and after its own slice."""
    first, count = shard_range(n, rank, world_size)
    return [(0, first), (first + count, n - first - count)]


@dataclass(frozen=True)
class ShardPlan:
    n: int
    rank: int
    world_size: int

    @property
    def first(self) -> int:
        return shard_range(self.n, self.rank, self.world_size)[0]

    @property
    def count(self) -> int:
        return shard_range(self.n, self.rank, self.world_size)[1]

    def active_slice(self, num_active: int) -> tuple[int, int]:
        """(first, count) of this rank's bodies that advance for a given num_active."""
        hi = min(self.first + self.count, active_bodies(num_active, self.n))
        return self.first, max(0, hi - self.first)

    def gather_bytes_sent(self) -> int:
        return 16 * self.count

    def gather_bytes_received(self) -> int:
        return 16 * (self.n - self.count)


# ---------------------------------------------------------------------------------------------
# Host-side mirrors of two device-side index maps, so that their combinatorics are checked on a
# CPU (tests/test_sym_cpu.py); the arithmetic itself lives in csrc/mapn_sym.hip (force_sym_kernel,
# sym_reduce_integrate_kernel) and csrc/mapn_kernels.hip / mapn_context.cpp (chunk_tiles, flow_row).

def sym_meetings(nb: int):
    """Meetings of the symmetric kernel: yields (a, partner block, d, symmetric) for every I-block
    a of `nb`, 16 J-blocks of 64 bodies each (not expanded here).  d = 0: the block against itself, one-sided;
    1 <= d <= D = (nb-1)//2: partner a+d, symmetric; for even nb also d = nb/2 when a is the RUNNER of the half-ring pair
    (sym_runs_half: the pairs alternate between the two halves of the ring)."""
    D = (nb - 1) // 2
    half = nb // 2 if nb % 2 == 0 else 0
    for a in range(nb):
        yield a, a, 0, False
        for d in range(1, D + 1):
            yield a, (a + d) % nb, d, True
        if sym_runs_half(nb, a):
            yield a, (a + half) % nb, half, True


def sym_runs_half(nb: int, a: int) -> bool:
    """csrc/mapn_kernels.h sym_runs_half: of the half-ring pair (p, p + nb/2) block p runs the meetings when p is even, block
    p + nb/2 when p is odd -- the extra group is spread evenly over the two halves of the ring (over the ranks of a sharded job)."""
    if nb % 2:
        return False
    half = nb // 2
    low = a < half
    p = a if low else a - half
    return (p % 2 == 0) == low


def sym_reaction_rows(nb: int, block: int) -> list[int]:
    """Groups g - 1 (g = 1 .. D, D + 1 = the half-ring group) whose rows the reduce kernel sums for a J-block lying in
    I-block `block`, over all windows of a step."""
    D = (nb - 1) // 2
    half = nb // 2 if nb % 2 == 0 else 0
    return list(range(D + (1 if half and not sym_runs_half(nb, block) else 0)))


def sym_wave_pieces(plan, window: int, set_: int):
    """How force_sym_kernel walks a block's meetings: yields (part, wave in the workgroup, meeting of the window, first
    step, steps) for every piece of every wave, from the host-built plan tables (`plan`: mapn.SymPlan -- the tables the
    kernel itself reads).  Wave v runs the linear steps [bounds[v], bounds[v + 1]); step 64 m + k is step k of meeting m."""
    b = plan.bounds(window, set_)      # set_: the block's class, or class + 2 * (block mod 8) with XCD-weighted parts (plan.set_of)
    for v in range(plan.nwaves):
        t, t1 = int(b[v]), int(b[v + 1])
        while t < t1:
            k0 = t & 63
            n = min(64 - k0, t1 - t)
            yield v // plan.waves, v % plan.waves, t >> 6, k0, n
            t += n


def sym_block_class(nb: int, a: int) -> int:
    """0: the block also runs the half-ring group (even nb, the runner of its pair: sym_runs_half); 1: the others."""
    return 0 if sym_runs_half(nb, a) else 1


def sym_shard_masks(nb: int, world: int, rank: int, blocks_per_rank: int | None = None) -> tuple[int, int]:
    """Gather algorithm 4 (mapn_context.cpp sym_shard_masks): bit q of `send` = this rank produces
    reactions for bodies of rank q, bit q of `recv` = rank q produces reactions for this rank's bodies;
    blocks are owned in contiguous runs of nb // world (blocks_per_rank: the ring of the ACTIVE blocks of a partially active
    step -- nb of them, still owned in runs of a rank's slice, so the last ranks may own none)."""
    nbl = blocks_per_rank if blocks_per_rank is not None else nb // world
    send = recv = 0
    for a, b, d, symmetric in sym_meetings(nb):
        if not symmetric:
            continue
        if a // nbl == rank:
            send |= 1 << (b // nbl)
        if b // nbl == rank:
            recv |= 1 << (a // nbl)
    return send, recv


def chunk_tiles(tiles: int, splits: int, chunk: int) -> tuple[int, int]:
    """64-body tiles [t0, t1) of chunk `chunk` of `splits` (mapn_kernels.hip chunk_tiles)."""
    base, rem = divmod(tiles, splits)
    t0 = chunk * base + min(chunk, rem)
    return t0, t0 + base + (1 if chunk < rem else 0)


def flow_row_rotation(n: int, first: int, waves: int, sb: int) -> int:
    """Block-row rotation of flow mode (mapn_context.cpp enqueue_step): the row holding the first
    tile of the rank's own slice is dispatched first."""
    tiles = (n + 63) // 64
    base, rem = divmod(tiles, waves * sb)
    t_own = first // 64
    if t_own < rem * (base + 1):
        c_own = t_own // (base + 1)
    else:
        c_own = rem + (t_own - rem * (base + 1)) // base if base else 0
    return min(c_own // waves, sb - 1)
