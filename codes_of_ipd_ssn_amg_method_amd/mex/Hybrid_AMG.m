function [zeta,itamg,resamg,info] = Hybrid_AMG(prob_data,amg_options)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[zeta,itamg,resamg,info] = ipd_mex('Hybrid_AMG', prob_data,amg_options);
end
