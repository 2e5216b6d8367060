"""Frame partitioning of the data-parallel path: the reference's DistributedGroupSampler
(det3d/datasets/loader/sampler.py:138-224), same arithmetic and the same torch.Generator stream, so for a given
(epoch, flags, samples_per_gpu, world size) every rank draws exactly the indices the reference would:
  per aspect-ratio group: epoch-seeded permutation, padded with its own head to a multiple of
  samples_per_gpu * num_replicas;  then a permutation of whole samples_per_gpu-sized chunks;  rank r takes the
  contiguous slice [r * num_samples, (r + 1) * num_samples).
"""
import math

import numpy as np
import torch
from torch.utils.data import Sampler

from . import dist as d3d_dist


def group_num_samples(flag, samples_per_gpu, num_replicas):
    sizes = np.bincount(np.asarray(flag))
    return sum(int(math.ceil(s * 1.0 / samples_per_gpu / num_replicas)) * samples_per_gpu for s in sizes)


def partition_indices(flag, samples_per_gpu, num_replicas, rank, epoch):
    flag = np.asarray(flag)
    sizes = np.bincount(flag)
    num_samples = group_num_samples(flag, samples_per_gpu, num_replicas)
    g = torch.Generator()
    g.manual_seed(epoch)
    indices = []
    for i, size in enumerate(sizes):
        if size > 0:
            ind = np.where(flag == i)[0]
            ind = ind[torch.randperm(int(size), generator=g).tolist()].tolist()
            total = int(math.ceil(size * 1.0 / samples_per_gpu / num_replicas)) * samples_per_gpu * num_replicas
            ind += ind[: total - len(ind)]
            indices += ind
    assert len(indices) == num_samples * num_replicas
    chunks = torch.randperm(len(indices) // samples_per_gpu, generator=g).tolist()
    indices = [indices[j] for c in chunks for j in range(c * samples_per_gpu, (c + 1) * samples_per_gpu)]
    return indices[num_samples * rank: num_samples * (rank + 1)]


class DistributedGroupSampler(Sampler):
    """Same constructor and behaviour as the reference class; `dataset` needs a `.flag` array."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        self.dataset = dataset
        self.samples_per_gpu = samples_per_gpu
        self.num_replicas = d3d_dist.world_size() if num_replicas is None else num_replicas
        self.rank = d3d_dist.get_rank() if rank is None else rank
        self.epoch = 0
        assert hasattr(dataset, "flag")
        self.flag = np.asarray(dataset.flag)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = group_num_samples(self.flag, samples_per_gpu, self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        return iter(partition_indices(self.flag, self.samples_per_gpu, self.num_replicas, self.rank, self.epoch))

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
