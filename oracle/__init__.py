"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (pure PyTorch-CPU + numpy/scipy) of the PointNet2 hot path of
IGNF/StrataNet2-Vegetation-Coverage-Maps: `model/point_net2.py` (SA / FP stacks, head) and
`model/project_to_2d.py` (max-projection rasters), plus the third-party primitives those files call
(torch-cluster 1.5.9 `fps`/`radius`/`knn`, torch-geometric 1.7.2 `PointConv`/`knn_interpolate`/
`global_max_pool`, torch-scatter 2.0.7 `scatter_max`/`scatter_mean`; pins:
`setup_environment/torch_extensions.txt:1-3`), none of which are installed here.

Who may import this package: `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg --
as the checker / the timed CPU baseline, never as the product path.  The product
(`stratanet2_vegetation_coverage_maps_amd`) never imports it and fails loudly without its HIP library.

Parity status
-------------
* Reference glue (`PointNet2.forward`, `SAModule`, `GlobalSAModule`, `FPModule`, `MLP`,
  `project_to_plotwise_coverages`, `project_to_2d_rasters`, `loss_functions`): PINNED -- the goldens under
  `tests/golden/` were produced by `oracle/make_golden.py`, which imports that code from
  `/root/reference` in the build container and runs it; `oracle/network.py`, `oracle/projection.py` and
  `oracle/losses.py` are checked against those goldens by `tests/test_oracle_golden.py`.
* Rows next to the hot path -- PINNED by `oracle/make_golden_aux.py`, which imports and RUNS the
  reference's own numpy code in the build container and writes `tests/golden/f_*.npz`:
    - `oracle/mosaic.py` (weights band, geotransform, the rasterio.merge callback in float32, hard medium-vegetation
      band, finalisation) vs `inference/geotiff_raster.py:46-61,103-144,262-347`: bit for bit;
    - `oracle/prepare.py::load_cloud` vs `data_loader/loader.py:73-255` under the same `numpy.random` seed: bit for bit
      (also recorded in the fixture: the numpy-1.21-casting restatement and the loader under numpy 2.2 differ by 0.0);
    - `oracle/prepare.py::normalize_z_with_minz_in_a_radius` vs `utils/load_data.py:237-249` on the real sklearn
      kd-tree (points at exactly the radius, Lambert-sized coordinates): bit for bit.
  `tests/test_oracle_golden_aux.py` holds the restatements to those fixtures; the `-m gpu` tests hold the HIP kernels to
  the same fixtures.  Unpinned remainder: the rounding of a geotransform to integer pixel offsets (rasterio's job).
* `oracle/check.py`: one training step of the oracle in a chosen precision; the GPU tests use its fp64 evaluation as the
  yardstick (same fp32 geometry), because the fp32 CPU evaluation is itself ~1e-3 away from the exact network at the
  metric's size while the HIP path is ~2e-6 away (tests/test_gpu_metric_size.py prints both).
* Third-party primitives (`oracle/primitives.py`): PARITY UNPINNED -- the reference holds no tests,
  fixtures or golden vectors and the wheels are absent and un-fetchable, so the
  primitives are restated from their published algorithms; every assumption is listed in the
  docstring of the function that makes it, and `tests/test_oracle_primitives.py` pins the restatement to
  hand-computed known answers and to scipy's kd-tree on exactly those assumptions.
"""
