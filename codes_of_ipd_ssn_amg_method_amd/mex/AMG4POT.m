function [zeta,itamg,resamg,info] = AMG4POT(prob_data,amg_options,str)
% Drop-in shim (Class2/AMG4POT.m:1).  Only the 'amg' inner solver is built.
if ~strcmp(str,'amg'), error('ipdamg:unsupported','only the ''amg'' inner solver is built'); end
[zeta,itamg,resamg,info] = ipd_mex('AMG4POT', prob_data, amg_options);
end
