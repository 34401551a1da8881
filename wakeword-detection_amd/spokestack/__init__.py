"""Import-path compatibility: ``spokestack.*`` names of the reference resolve to the HIP-backed
implementations in ``wwhip`` (add ``wakeword-detection_amd/`` to ``sys.path``)."""
