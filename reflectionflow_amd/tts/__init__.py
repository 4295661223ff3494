"""Search drivers of the test-time-scaling loop (mirror of the reference's `tts/` scripts for the
denoise path): noise protocol, candidate sharding across the GPUs of a node, round-boundary
exchange of verifier scores."""
