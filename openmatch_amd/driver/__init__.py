"""Command-line entry points with the reference's module names and flags
(`python -m openmatch.driver.build_index|retrieve|successive_retrieve|train_dr|rerank`)."""
