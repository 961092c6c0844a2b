function [isC,isF,As] = mis_set(varargin)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[isC,isF,As] = ipd_mex('mis_set', varargin{:});
end
