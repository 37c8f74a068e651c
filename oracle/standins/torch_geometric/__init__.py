"""Stand-in for the slice of torch-geometric 2.1.0.post1 the reference's hot path imports."""
