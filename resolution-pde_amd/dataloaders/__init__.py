"""On-disk formats -> Markov pairs.  Host-side; feeds the HIP hot path."""
