function [d,it,res,resk] = PCG(varargin)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[d,it,res,resk] = ipd_mex('PCG', varargin{:});
end
