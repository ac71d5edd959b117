"""Host-side mirror of the reference's `nerfactor` package surface (models / networks / util) on
PyTorch-ROCm + libnfx.  Put `<repo>/nerfactor_amd` on PYTHONPATH the way the reference's run scripts
put its repo root there (nerfactor/trainvali_run.sh:29-33) and `import nerfactor.models.nerf`,
`import brdf.renderer` resolve to this implementation."""
