"""Body sharding across ranks (one process per GPU).

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
