function [zeta,itamg,resamg,info] = Hybrid_twogrid(prob_data,amg_options)
% Drop-in shim (Hybrid_twogrid.m:1); forwards to libipdamg.
[zeta,itamg,resamg,info] = ipd_mex('Hybrid_twogrid', prob_data, amg_options);
end
