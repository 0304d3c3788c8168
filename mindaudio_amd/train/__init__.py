"""Training step of the Conformer CTC path on MI355X: hand-written backward kernels, Adam with the
reference's dynamic loss scale, and RCCL data parallelism."""
