"""tedd: the time-stepping hot path of the temperature-enhanced drug-delivery diffusion model on
MI355X.  `tedd.prob1base`, `tedd.prob1_mms_cases`, `tedd.mms_trial_utils` and
`tedd.cvg_studies_base` mirror the reference's modules of the same names; `tedd._ffi` binds the
C ABI of libtedd_hip.so; `tedd.distributed` splits the mesh into row slabs across GPUs."""
