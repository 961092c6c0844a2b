function H = ASAt(s,p,q)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[H] = ipd_mex('ASAt', s,p,q);
end
