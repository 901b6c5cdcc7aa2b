"""Trainer-side fused operators: what sits between MIOpen's convolutions in the ResNet-18 trunk."""
