function [zeta,itpcg,respcg,info] = PCG4POT(prob_data,pcg_options)
% Drop-in shim (Class2/PCG4POT.m:1); forwards to libipdamg.
[zeta,itpcg,respcg,info] = ipd_mex('PCG4POT', prob_data, pcg_options);
end
