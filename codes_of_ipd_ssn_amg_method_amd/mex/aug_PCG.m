function [zeta,itpcg,respcg,info] = aug_PCG(prob_data,pcg_options)
% Drop-in shim (aug_PCG.m:1); forwards to libipdamg.
[zeta,itpcg,respcg,info] = ipd_mex('aug_PCG', prob_data, pcg_options);
end
