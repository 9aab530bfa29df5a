"""Modules behind bench.py (the CLI and the dispatch stay there): common (constants, process group, launcher, profile readers),
frame (the headline line + weak emulation), strong (one frame partitioned over the ranks + its emulation), train, roofline,
baselines (CPU oracle / torch / eager-GPU / host-to-host), dry (launcher + collectives on the CPU)."""
