"""Drop-in for the inference-only model copies of v2/downstream: the same towers on the HIP engine
with no tube masking, no transcript-sorting head and no `args` constructor argument."""
