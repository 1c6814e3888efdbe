"""Data-parallel collectives (RCCL through torch.distributed), gradient arena, training utilities."""
