"""Per-tensor-kind Omega / Psi contractions (device implementations of the reference's
``tt_sketch/sketching_methods``), consuming the output of the matching ``DRM.sketch_<kind>``."""
