"""Model description front-ends (build-time only)."""
