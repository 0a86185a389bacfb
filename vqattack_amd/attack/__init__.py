"""Attack drivers above the operator API: block scheduler, text substitution, ASR bookkeeping, rank sharding."""
